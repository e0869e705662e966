// rm_addon.cc -- N-API binding of the C ABI (include/rm_raymarch.h) for a Node / Electron /
// worker_threads host of the reference's TypeScript.  It is the binding INTEGRATION.md
// describes: the worker keeps `Job` / `Result` (src/workers/raymarchWorker.ts:10-31) and
// replaces the body of onmessage (raymarchWorker.ts:33-92) with renderTile(job, buffers).
//
// One rm_ctx per addon INSTANCE, i.e. per napi_env: node loads the addon once per worker_threads
// worker (the .so itself is mapped once per process), so each worker of the reference's pool
// (main.ts:318-321) owns its own context.  The ABI is not thread-safe per ctx; nothing here is
// process-global.  The ctx lives in the env's instance data and is destroyed when the env goes.
// Nothing here throws: every call returns the rm_status code (0 ok, -2 = RM_E_UNSUPPORTED ->
// the caller keeps its CPU path for that job), lastError() has the text.
#define NAPI_VERSION 6  // napi_set_instance_data (node >= 12.17)
#include <node_api.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>

#include "../include/rm_raymarch.h"

namespace {

struct Addon {
    rm_ctx *ctx = nullptr;
};

void addon_finalize(napi_env, void *data, void *) {
    Addon *a = static_cast<Addon *>(data);
    if (a->ctx) rm_destroy(a->ctx);
    delete a;
}

Addon *addon_of(napi_env env) {
    void *data = nullptr;
    if (napi_get_instance_data(env, &data) != napi_ok) return nullptr;
    return static_cast<Addon *>(data);
}

double num_prop(napi_env env, napi_value obj, const char *name, double dflt) {
    napi_value v;
    bool has = false;
    if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return dflt;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok) return dflt;
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_number) return dflt;
    double d = dflt;
    napi_get_value_double(env, v, &d);
    return d;
}

std::string str_prop(napi_env env, napi_value obj, const char *name) {
    napi_value v;
    bool has = false;
    if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return "";
    if (napi_get_named_property(env, obj, name, &v) != napi_ok) return "";
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_string) return "";
    char buf[64];
    size_t n = 0;
    napi_get_value_string_utf8(env, v, buf, sizeof buf, &n);
    return std::string(buf, n);
}

napi_value obj_prop(napi_env env, napi_value obj, const char *name) {
    napi_value v = nullptr;
    bool has = false;
    if (napi_has_named_property(env, obj, name, &has) == napi_ok && has) napi_get_named_property(env, obj, name, &v);
    return v;
}

// Job (raymarchWorker.ts:10-22) -> rm_job
rm_job job_from_js(napi_env env, napi_value j) {
    rm_job job;
    std::memset(&job, 0, sizeof job);
    job.width = static_cast<int32_t>(num_prop(env, j, "width", 0));
    job.height = static_cast<int32_t>(num_prop(env, j, "height", 0));
    job.time = num_prop(env, j, "time", 0);
    job.y_start = static_cast<int32_t>(num_prop(env, j, "yStart", 0));
    job.y_end = static_cast<int32_t>(num_prop(env, j, "yEnd", job.height));
    napi_value cam = obj_prop(env, j, "camera");
    if (cam) {
        job.camera_pitch = num_prop(env, cam, "pitch", 0);
        job.camera_yaw = num_prop(env, cam, "yaw", 0);
    }
    job.algorithm = rm_algorithm_from_string(str_prop(env, j, "algorithm").c_str());
    job.scene_preset_index = static_cast<int32_t>(num_prop(env, j, "scenePresetIndex", 0));
    job.acceleration_structure = rm_accel_from_string(str_prop(env, j, "accelerationStructure").c_str());
    job.overshoot_factor = num_prop(env, j, "overshootFactor", NAN);  // undefined -> constructor default
    job.step_size = num_prop(env, j, "stepSize", NAN);
    return job;
}

// typed array -> data pointer + byte length (nullptr when it is not a typed array)
void *typed(napi_env env, napi_value v, size_t *bytes) {
    bool is = false;
    *bytes = 0;
    if (!v || napi_is_typedarray(env, v, &is) != napi_ok || !is) return nullptr;
    napi_typedarray_type t;
    size_t len = 0, off = 0;
    void *data = nullptr;
    napi_value ab;
    if (napi_get_typedarray_info(env, v, &t, &len, &data, &ab, &off) != napi_ok) return nullptr;
    const size_t esz = (t == napi_uint16_array || t == napi_int16_array) ? 2 : (t == napi_uint8_array || t == napi_uint8_clamped_array || t == napi_int8_array) ? 1 : 4;
    *bytes = len * esz;
    return data;
}

napi_value make_int(napi_env env, int v) {
    napi_value out;
    napi_create_int32(env, v, &out);
    return out;
}

// create(device) -> status
napi_value Create(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value a[1];
    napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    int32_t dev = 0;
    if (argc >= 1) napi_get_value_int32(env, a[0], &dev);
    Addon *ad = addon_of(env);
    if (!ad) return make_int(env, RM_E_INVALID);
    if (ad->ctx) {  // this env's own context only: another worker's context is never touched
        rm_destroy(ad->ctx);
        ad->ctx = nullptr;
    }
    return make_int(env, rm_create(dev, &ad->ctx));
}

// renderTile(job, depth: Uint8ClampedArray, normal: Uint8ClampedArray, sdfEval: Uint16Array, iters: Uint16Array)
napi_value RenderTile(napi_env env, napi_callback_info info) {
    size_t argc = 5;
    napi_value a[5];
    napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    Addon *ad = addon_of(env);
    rm_ctx *g_ctx = ad ? ad->ctx : nullptr;
    if (!g_ctx || argc < 5) return make_int(env, RM_E_INVALID);
    const rm_job job = job_from_js(env, a[0]);
    const size_t rows = job.y_end > job.y_start ? static_cast<size_t>(job.y_end - job.y_start) : 0;
    const size_t npx = rows * static_cast<size_t>(job.width > 0 ? job.width : 0);
    size_t b0, b1, b2, b3;
    uint8_t *depth = static_cast<uint8_t *>(typed(env, a[1], &b0));
    uint8_t *normal = static_cast<uint8_t *>(typed(env, a[2], &b1));
    uint16_t *sdf = static_cast<uint16_t *>(typed(env, a[3], &b2));
    uint16_t *iters = static_cast<uint16_t *>(typed(env, a[4], &b3));
    if (b0 < npx || b1 < 3 * npx || b2 < 2 * npx || b3 < 2 * npx) return make_int(env, RM_E_INVALID);
    return make_int(env, rm_render_tile(g_ctx, &job, depth, normal, sdf, iters));
}

// shade(modelName, width, height, depth, normal, sdfEval, iters, rgba)   (shadingModel.ts:8-17)
napi_value Shade(napi_env env, napi_callback_info info) {
    size_t argc = 8;
    napi_value a[8];
    napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    Addon *ad = addon_of(env);
    rm_ctx *g_ctx = ad ? ad->ctx : nullptr;
    if (!g_ctx || argc < 8) return make_int(env, RM_E_INVALID);
    char name[64];
    size_t n = 0;
    napi_get_value_string_utf8(env, a[0], name, sizeof name, &n);
    int32_t w = 0, h = 0;
    napi_get_value_int32(env, a[1], &w);
    napi_get_value_int32(env, a[2], &h);
    const size_t npx = static_cast<size_t>(w > 0 ? w : 0) * static_cast<size_t>(h > 0 ? h : 0);
    size_t b[5];
    void *p[5];
    for (int i = 0; i < 5; ++i) p[i] = typed(env, a[3 + i], &b[i]);
    if (b[0] < npx || b[1] < 3 * npx || b[2] < 2 * npx || b[3] < 2 * npx || b[4] < 4 * npx) return make_int(env, RM_E_INVALID);
    return make_int(env, rm_shade(g_ctx, rm_shader_from_string(name), w, h, static_cast<uint8_t *>(p[0]),
                                  static_cast<uint8_t *>(p[1]), static_cast<uint16_t *>(p[2]),
                                  static_cast<uint16_t *>(p[3]), static_cast<uint8_t *>(p[4])));
}

// diagnostics(sdfEval, iters) -> {totalSDFCalls, maxSDFCalls, minSDFCalls, totalIterations, totalPixels} | status
napi_value Diagnostics(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value a[2];
    napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    Addon *ad = addon_of(env);
    rm_ctx *g_ctx = ad ? ad->ctx : nullptr;
    if (!g_ctx || argc < 2) return make_int(env, RM_E_INVALID);
    size_t b0, b1;
    uint16_t *sdf = static_cast<uint16_t *>(typed(env, a[0], &b0));
    uint16_t *iters = static_cast<uint16_t *>(typed(env, a[1], &b1));
    if (b0 != b1) return make_int(env, RM_E_INVALID);
    rm_diagnostics d;
    const int rc = rm_reduce_counters(g_ctx, sdf, iters, static_cast<int64_t>(b0 / 2), &d);
    if (rc) return make_int(env, rc);
    napi_value out, v;
    napi_create_object(env, &out);
    napi_create_double(env, static_cast<double>(d.total_sdf_calls), &v);
    napi_set_named_property(env, out, "totalSDFCalls", v);
    napi_create_double(env, static_cast<double>(d.max_sdf_calls), &v);
    napi_set_named_property(env, out, "maxSDFCalls", v);
    napi_create_double(env, static_cast<double>(d.min_sdf_calls), &v);
    napi_set_named_property(env, out, "minSDFCalls", v);
    napi_create_double(env, static_cast<double>(d.total_iterations), &v);
    napi_set_named_property(env, out, "totalIterations", v);
    napi_create_double(env, static_cast<double>(d.total_pixels), &v);
    napi_set_named_property(env, out, "totalPixels", v);
    return out;
}

napi_value LastError(napi_env env, napi_callback_info) {
    napi_value out;
    Addon *ad = addon_of(env);
    const char *s = rm_last_error(ad ? ad->ctx : nullptr);
    napi_create_string_utf8(env, s, std::strlen(s), &out);
    return out;
}

napi_value Version(napi_env env, napi_callback_info) {
    napi_value out;
    const char *s = rm_version();
    napi_create_string_utf8(env, s, std::strlen(s), &out);
    return out;
}

napi_value Init(napi_env env, napi_value exports) {
    napi_set_instance_data(env, new Addon(), addon_finalize, nullptr);
    const napi_property_descriptor props[] = {
        {"create", nullptr, Create, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"renderTile", nullptr, RenderTile, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"shade", nullptr, Shade, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"diagnostics", nullptr, Diagnostics, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"lastError", nullptr, LastError, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"version", nullptr, Version, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    };
    napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
    return exports;
}

}  // namespace

NAPI_MODULE(rm_addon, Init)
