/*
 * N-API binding of the C ABI in include/rt_hip.h: one JavaScript function per entry point, typed arrays in,
 * typed arrays out, no logic. Structs (rt_config, rt_prim, rt_material, rt_light) are packed on the JavaScript
 * side into ArrayBuffers with the C layout (structSizes() lets it assert the sizes) and passed through.
 * A failing entry point becomes a thrown Error carrying rt_last_error().
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_hip.h"

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

/* RT_ADDON_BACKTRACE=1: print the native stack on SIGSEGV before dying (diagnostic aid, off by default) */
static void on_segv(int sig) {
    void *frames[64];
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

#define NAPI_OK(call)                                                            \
    do {                                                                         \
        if ((call) != napi_ok) {                                                 \
            napi_throw_error(env, NULL, "rt_addon: N-API call failed: " #call);  \
            return NULL;                                                         \
        }                                                                        \
    } while (0)

#define ARGS(n)                                                      \
    size_t argc = (n);                                               \
    napi_value argv[8];                                              \
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));   \
    if (argc < (n)) {                                                \
        napi_throw_type_error(env, NULL, "rt_addon: too few arguments"); \
        return NULL;                                                 \
    }

static napi_value throw_rt(napi_env env, rt_ctx *ctx, const char *what) {
    char msg[512];
    snprintf(msg, sizeof msg, "%s: %s", what, rt_last_error(ctx));
    napi_throw_error(env, NULL, msg);
    return NULL;
}

/* Context handles are externals WITHOUT a garbage-collection finalizer: Node 12 can run N-API finalizers during
 * environment teardown after the structures they need are gone (seen as a segmentation fault at process exit
 * inside libnode). Instead every live context sits in a list that an environment cleanup hook empties, and
 * RayTracer.dispose() releases a context early. A handle that is dropped without dispose() lives until exit. */
typedef struct ctx_slot {
    rt_ctx *ctx;
    struct ctx_slot *next;
} ctx_slot;
static ctx_slot *g_slots = NULL;

static void cleanup_all(void *arg) {
    (void)arg;
    for (ctx_slot *s = g_slots; s; s = s->next) {
        if (s->ctx) rt_destroy(s->ctx);
        s->ctx = NULL;
    }
}

static int get_ctx(napi_env env, napi_value v, rt_ctx **out) {
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "rt_addon: first argument must be a live context handle");
        return 0;
    }
    ctx_slot *slot = (ctx_slot *)p;
    if (!slot->ctx) {
        napi_throw_error(env, NULL, "rt_addon: context was destroyed");
        return 0;
    }
    *out = slot->ctx;
    return 1;
}

/* data pointer + byte length of an ArrayBuffer or of any TypedArray / DataView; null/undefined -> (NULL, 0) */
static int get_bytes(napi_env env, napi_value v, void **data, size_t *bytes) {
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok) return 0;
    if (t == napi_null || t == napi_undefined) { *data = NULL; *bytes = 0; return 1; }
    bool is = false;
    if (napi_is_arraybuffer(env, v, &is) == napi_ok && is) return napi_get_arraybuffer_info(env, v, data, bytes) == napi_ok;
    if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
        napi_typedarray_type ty;
        size_t len, off;
        napi_value ab;
        if (napi_get_typedarray_info(env, v, &ty, &len, data, &ab, &off) != napi_ok) return 0;
        static const size_t width[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
        *bytes = len * width[ty];
        return 1;
    }
    if (napi_is_dataview(env, v, &is) == napi_ok && is) {
        napi_value ab;
        size_t off;
        return napi_get_dataview_info(env, v, bytes, data, &ab, &off) == napi_ok;
    }
    napi_throw_type_error(env, NULL, "rt_addon: expected an ArrayBuffer or a typed array");
    return 0;
}

/* ---- life cycle --------------------------------------------------------------------------------- */
static napi_value js_struct_sizes(napi_env env, napi_callback_info info) {
    (void)info;
    napi_value o, v;
    NAPI_OK(napi_create_object(env, &o));
#define PUT(name, val) NAPI_OK(napi_create_uint32(env, (uint32_t)(val), &v)); NAPI_OK(napi_set_named_property(env, o, name, v));
    PUT("config", sizeof(rt_config)) PUT("prim", sizeof(rt_prim)) PUT("material", sizeof(rt_material))
    PUT("light", sizeof(rt_light)) PUT("stats", sizeof(rt_stats))
#undef PUT
    return o;
}

static napi_value js_default_config(napi_env env, napi_callback_info info) {
    (void)info;
    void *p;
    napi_value ab;
    NAPI_OK(napi_create_arraybuffer(env, sizeof(rt_config), &p, &ab));
    rt_default_config((rt_config *)p);
    return ab;
}

static napi_value js_create(napi_env env, napi_callback_info info) {
    ARGS(1)
    void *p; size_t n;
    if (!get_bytes(env, argv[0], &p, &n)) return NULL;
    if (n != sizeof(rt_config)) { napi_throw_type_error(env, NULL, "rt_create: config buffer has the wrong size"); return NULL; }
    rt_ctx *ctx = NULL;
    if (rt_create((const rt_config *)p, &ctx) != 0) return throw_rt(env, NULL, "rt_create");
    ctx_slot *slot = (ctx_slot *)malloc(sizeof *slot);   /* never freed: a stale handle must stay readable */
    if (!slot) { rt_destroy(ctx); napi_throw_error(env, NULL, "rt_create: out of memory"); return NULL; }
    slot->ctx = ctx;
    slot->next = g_slots;
    g_slots = slot;
    napi_value ext;
    NAPI_OK(napi_create_external(env, slot, NULL, NULL, &ext));
    return ext;
}

static napi_value js_destroy(napi_env env, napi_callback_info info) {
    ARGS(1)
    void *p = NULL;
    if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
        ctx_slot *slot = (ctx_slot *)p;
        if (slot->ctx) rt_destroy(slot->ctx);
        slot->ctx = NULL;
    }
    return NULL;
}

/* ---- scene ---------------------------------------------------------------------------------------- */
#define SET_ARRAY(fn_name, rt_fn, T)                                                               \
    static napi_value fn_name(napi_env env, napi_callback_info info) {                             \
        ARGS(2)                                                                                    \
        rt_ctx *ctx; void *p; size_t n;                                                            \
        if (!get_ctx(env, argv[0], &ctx) || !get_bytes(env, argv[1], &p, &n)) return NULL;         \
        if (n % sizeof(T)) { napi_throw_type_error(env, NULL, #rt_fn ": buffer is not a whole number of records"); return NULL; } \
        if (rt_fn(ctx, (const T *)p, (int32_t)(n / sizeof(T))) != 0) return throw_rt(env, ctx, #rt_fn); \
        return NULL;                                                                               \
    }
SET_ARRAY(js_set_primitives, rt_set_primitives, rt_prim)
SET_ARRAY(js_set_materials, rt_set_materials, rt_material)

static napi_value js_set_lights(napi_env env, napi_callback_info info) {
    ARGS(3)
    rt_ctx *ctx; void *p; size_t n; int32_t n_loop;
    if (!get_ctx(env, argv[0], &ctx) || !get_bytes(env, argv[1], &p, &n)) return NULL;
    NAPI_OK(napi_get_value_int32(env, argv[2], &n_loop));
    if (n % sizeof(rt_light)) { napi_throw_type_error(env, NULL, "rt_set_lights: buffer is not a whole number of records"); return NULL; }
    if (rt_set_lights(ctx, (const rt_light *)p, (int32_t)(n / sizeof(rt_light)), n_loop) != 0) return throw_rt(env, ctx, "rt_set_lights");
    return NULL;
}

static napi_value js_set_globals(napi_env env, napi_callback_info info) {
    ARGS(3)
    rt_ctx *ctx; void *a, *b; size_t na, nb;
    if (!get_ctx(env, argv[0], &ctx) || !get_bytes(env, argv[1], &a, &na) || !get_bytes(env, argv[2], &b, &nb)) return NULL;
    if (na != 12 || nb != 12) { napi_throw_type_error(env, NULL, "rt_set_globals: expected two Float32Array(3)"); return NULL; }
    if (rt_set_globals(ctx, (const float *)a, (const float *)b) != 0) return throw_rt(env, ctx, "rt_set_globals");
    return NULL;
}

static napi_value js_set_camera(napi_env env, napi_callback_info info) {
    ARGS(2)
    rt_ctx *ctx; void *p; size_t n;
    if (!get_ctx(env, argv[0], &ctx) || !get_bytes(env, argv[1], &p, &n)) return NULL;
    if (n != 15 * sizeof(float)) { napi_throw_type_error(env, NULL, "rt_set_camera: expected Float32Array(15) = pos, ray00, ray01, ray10, ray11"); return NULL; }
    const float *f = (const float *)p;
    if (rt_set_camera(ctx, f, f + 3, f + 6, f + 9, f + 12) != 0) return throw_rt(env, ctx, "rt_set_camera");
    return NULL;
}

static napi_value js_set_mesh(napi_env env, napi_callback_info info) {
    ARGS(4)
    rt_ctx *ctx; void *v, *f, *m; size_t nv, nf, nm;
    if (!get_ctx(env, argv[0], &ctx) || !get_bytes(env, argv[1], &v, &nv) || !get_bytes(env, argv[2], &f, &nf) ||
        !get_bytes(env, argv[3], &m, &nm)) return NULL;
    if (nv % 24 || nf % 12 || nm != nf / 12 * 8) { napi_throw_type_error(env, NULL, "rt_set_mesh: expected Float64Array(3*nV), Uint32Array(3*nF), Uint32Array(2*nF)"); return NULL; }
    if (rt_set_mesh(ctx, (const double *)v, (int32_t)(nv / 24), (const uint32_t *)f, (const uint32_t *)m, (int32_t)(nf / 12)) != 0)
        return throw_rt(env, ctx, "rt_set_mesh");
    return NULL;
}

/* rt_update_vertices: moved vertices under the tree as it stands (the frames between two k-d rebuilds) */
static napi_value js_update_vertices(napi_env env, napi_callback_info info) {
    ARGS(2)
    rt_ctx *ctx; void *v; size_t nv;
    if (!get_ctx(env, argv[0], &ctx) || !get_bytes(env, argv[1], &v, &nv)) return NULL;
    if (nv % 24) { napi_throw_type_error(env, NULL, "rt_update_vertices: expected Float64Array(3*nV)"); return NULL; }
    if (rt_update_vertices(ctx, (const double *)v, (int32_t)(nv / 24)) != 0) return throw_rt(env, ctx, "rt_update_vertices");
    return NULL;
}

#define CALL0(fn_name, rt_fn)                                            \
    static napi_value fn_name(napi_env env, napi_callback_info info) {   \
        ARGS(1)                                                          \
        rt_ctx *ctx;                                                     \
        if (!get_ctx(env, argv[0], &ctx)) return NULL;                   \
        if (rt_fn(ctx) != 0) return throw_rt(env, ctx, #rt_fn);          \
        return NULL;                                                     \
    }
CALL0(js_build_kdtree, rt_build_kdtree)
CALL0(js_build_kdtree_device, rt_build_kdtree_device)
CALL0(js_trace, rt_trace)
CALL0(js_trace_begin, rt_trace_begin)

/* rt_trace_end: true when the frame had to be traced again with larger buffers */
static napi_value js_trace_end(napi_env env, napi_callback_info info) {
    ARGS(1)
    rt_ctx *ctx;
    if (!get_ctx(env, argv[0], &ctx)) return NULL;
    int rc = rt_trace_end(ctx);
    if (rc < 0) return throw_rt(env, ctx, "rt_trace_end");
    napi_value b;
    NAPI_OK(napi_get_boolean(env, rc == 1, &b));
    return b;
}

/* root.min, root.max of the current tree into a Float64Array(6), without exporting the tree */
static napi_value js_kd_root_box(napi_env env, napi_callback_info info) {
    ARGS(2)
    rt_ctx *ctx; void *b; size_t nb;
    if (!get_ctx(env, argv[0], &ctx) || !get_bytes(env, argv[1], &b, &nb)) return NULL;
    if (nb != 48) { napi_throw_type_error(env, NULL, "kdRootBox: expected Float64Array(6)"); return NULL; }
    if (rt_export_kdtree_reference(ctx, NULL, NULL, (double *)b) != 0) return throw_rt(env, ctx, "rt_export_kdtree_reference");
    return NULL;
}

static napi_value js_kdtree_counts(napi_env env, napi_callback_info info) {
    ARGS(1)
    rt_ctx *ctx;
    if (!get_ctx(env, argv[0], &ctx)) return NULL;
    int64_t nn = 0, nl = 0;
    if (rt_kdtree_counts(ctx, &nn, &nl) != 0) return throw_rt(env, ctx, "rt_kdtree_counts");
    napi_value arr, a, b;
    NAPI_OK(napi_create_array_with_length(env, 2, &arr));
    NAPI_OK(napi_create_double(env, (double)nn, &a));
    NAPI_OK(napi_create_double(env, (double)nl, &b));
    NAPI_OK(napi_set_element(env, arr, 0, a));
    NAPI_OK(napi_set_element(env, arr, 1, b));
    return arr;
}

static napi_value js_export_kdtree(napi_env env, napi_callback_info info) {
    ARGS(4)
    rt_ctx *ctx; void *n, *l, *b; size_t nn, nl, nb;
    if (!get_ctx(env, argv[0], &ctx) || !get_bytes(env, argv[1], &n, &nn) || !get_bytes(env, argv[2], &l, &nl) ||
        !get_bytes(env, argv[3], &b, &nb)) return NULL;
    int64_t cn = 0, cl = 0;
    if (rt_kdtree_counts(ctx, &cn, &cl) != 0) return throw_rt(env, ctx, "rt_kdtree_counts");
    if (nn != (size_t)cn * 8 || nl != (size_t)cl * 16 || nb != 48) {
        napi_throw_type_error(env, NULL, "rt_export_kdtree_reference: expected Uint32Array(2*nodes), Int32Array(4*leaves), Float64Array(6)");
        return NULL;
    }
    if (rt_export_kdtree_reference(ctx, (uint32_t *)n, (int32_t *)l, (double *)b) != 0) return throw_rt(env, ctx, "rt_export_kdtree_reference");
    return NULL;
}

/* ---- frame ---------------------------------------------------------------------------------------- */

#define READ_INTO(fn_name, rt_fn, T)                                                         \
    static napi_value fn_name(napi_env env, napi_callback_info info) {                       \
        ARGS(3)                                                                              \
        rt_ctx *ctx; void *p; size_t n; uint32_t need;                                       \
        if (!get_ctx(env, argv[0], &ctx) || !get_bytes(env, argv[1], &p, &n)) return NULL;   \
        NAPI_OK(napi_get_value_uint32(env, argv[2], &need));                                 \
        if (n != (size_t)need) { napi_throw_type_error(env, NULL, #rt_fn ": output buffer has the wrong size"); return NULL; } \
        if (rt_fn(ctx, (T *)p) != 0) return throw_rt(env, ctx, #rt_fn);                      \
        return NULL;                                                                         \
    }
/* third argument = the byte size the JavaScript side computed from its own config; the library knows the true
 * size, the check guards against a caller passing a short buffer */
READ_INTO(js_read_float, rt_read_float, float)
READ_INTO(js_read_rgba8, rt_read_rgba8, uint8_t)
READ_INTO(js_read_hit_ids, rt_read_hit_ids, int32_t)

static napi_value js_pick(napi_env env, napi_callback_info info) {
    ARGS(3)
    rt_ctx *ctx; int32_t x, y, id = -1;
    if (!get_ctx(env, argv[0], &ctx)) return NULL;
    NAPI_OK(napi_get_value_int32(env, argv[1], &x));
    NAPI_OK(napi_get_value_int32(env, argv[2], &y));
    if (rt_pick(ctx, x, y, &id) != 0) return throw_rt(env, ctx, "rt_pick");
    napi_value v;
    NAPI_OK(napi_create_int32(env, id, &v));
    return v;
}

static napi_value js_tone(napi_env env, napi_callback_info info) {
    ARGS(1)
    rt_ctx *ctx;
    if (!get_ctx(env, argv[0], &ctx)) return NULL;
    double Lbar = 0, sf = 0;
    if (rt_tone(ctx, &Lbar, &sf) != 0) return throw_rt(env, ctx, "rt_tone");
    napi_value arr, a, b;
    NAPI_OK(napi_create_array_with_length(env, 2, &arr));
    NAPI_OK(napi_create_double(env, Lbar, &a));
    NAPI_OK(napi_create_double(env, sf, &b));
    NAPI_OK(napi_set_element(env, arr, 0, a));
    NAPI_OK(napi_set_element(env, arr, 1, b));
    return arr;
}

static napi_value js_present(napi_env env, napi_callback_info info) {
    ARGS(3)
    rt_ctx *ctx; void *p; size_t n; uint32_t need;
    if (!get_ctx(env, argv[0], &ctx) || !get_bytes(env, argv[1], &p, &n)) return NULL;
    NAPI_OK(napi_get_value_uint32(env, argv[2], &need));
    if (p && n != (size_t)need) { napi_throw_type_error(env, NULL, "rt_present: canvas buffer has the wrong size"); return NULL; }
    if (rt_present(ctx, (uint8_t *)p) != 0) return throw_rt(env, ctx, "rt_present");
    return NULL;
}

static napi_value js_stats(napi_env env, napi_callback_info info) {
    ARGS(1)
    rt_ctx *ctx;
    if (!get_ctx(env, argv[0], &ctx)) return NULL;
    rt_stats st;
    if (rt_get_stats(ctx, &st) != 0) return throw_rt(env, ctx, "rt_get_stats");
    napi_value o, v;
    NAPI_OK(napi_create_object(env, &o));
#define PUT(name, val) NAPI_OK(napi_create_double(env, (double)(val), &v)); NAPI_OK(napi_set_named_property(env, o, name, v));
    PUT("raysTotal", st.rays_total) PUT("shadowRays", st.shadow_rays) PUT("pixels", st.pixels) PUT("traceMs", st.trace_ms)
    PUT("compositeMs", st.composite_ms) PUT("toneMs", st.tone_ms) PUT("kdBuildMs", st.kd_build_ms) PUT("kdUploadMs", st.kd_upload_ms)
    PUT("kdClosestRays", st.kd_closest_rays) PUT("kdShadowRays", st.kd_shadow_rays) PUT("kdDeviceFallbacks", st.kd_device_fallbacks)
#undef PUT
    napi_value arr;
    NAPI_OK(napi_create_array_with_length(env, 8, &arr));
    for (uint32_t i = 0; i < 8; i++) {
        NAPI_OK(napi_create_double(env, (double)st.rays_per_level[i], &v));
        NAPI_OK(napi_set_element(env, arr, i, v));
    }
    NAPI_OK(napi_set_named_property(env, o, "raysPerLevel", arr));
    return o;
}

static napi_value init(napi_env env, napi_value exports) {
    if (getenv("RT_ADDON_BACKTRACE")) signal(SIGSEGV, on_segv);
    if (napi_add_env_cleanup_hook(env, cleanup_all, NULL) != napi_ok) {
        napi_throw_error(env, NULL, "rt_addon: cannot register the cleanup hook");
        return NULL;
    }
    static const struct { const char *name; napi_callback fn; } table[] = {
        {"structSizes", js_struct_sizes}, {"defaultConfig", js_default_config}, {"create", js_create}, {"destroy", js_destroy},
        {"setPrimitives", js_set_primitives}, {"setMaterials", js_set_materials}, {"setLights", js_set_lights},
        {"setGlobals", js_set_globals}, {"setCamera", js_set_camera}, {"setMesh", js_set_mesh}, {"updateVertices", js_update_vertices},
        {"buildKdTree", js_build_kdtree}, {"buildKdTreeDevice", js_build_kdtree_device}, {"kdtreeCounts", js_kdtree_counts},
        {"exportKdTree", js_export_kdtree}, {"kdRootBox", js_kd_root_box},
        {"trace", js_trace}, {"traceBegin", js_trace_begin}, {"traceEnd", js_trace_end}, {"readFloat", js_read_float}, {"readRGBA8", js_read_rgba8}, {"readHitIds", js_read_hit_ids},
        {"pick", js_pick}, {"tone", js_tone}, {"present", js_present}, {"stats", js_stats},
    };
    for (size_t i = 0; i < sizeof table / sizeof table[0]; i++) {
        napi_value fn;
        if (napi_create_function(env, table[i].name, NAPI_AUTO_LENGTH, table[i].fn, NULL, &fn) != napi_ok ||
            napi_set_named_property(env, exports, table[i].name, fn) != napi_ok) {
            napi_throw_error(env, NULL, "rt_addon: cannot register functions");
            return NULL;
        }
    }
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
