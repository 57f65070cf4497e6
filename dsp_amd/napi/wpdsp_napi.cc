// wpdsp_napi.cc — thin Node N-API glue over the C ABI in include/wpdsp.h.
//
// Plain C N-API (node_api.h, N-API v4+), no node-addon-api.  Context-aware:
// the module has no global JS state, so it can be loaded in the main thread
// and in any number of worker_threads (the reference runs one module instance
// per Web Worker, src/mix-worker-thread.js).  Typed arrays are read in place
// (byte offsets of subarray views honoured, SharedArrayBuffer backing accepted);
// results are fresh Float32Arrays with their own ArrayBuffer, like the
// reference's (lib/convolve.js:575,582).  Errors are thrown with the class the
// reference throws: RangeError for an over-long input (lib/convolve.js:590),
// Error for a bad FFT size (:8-9).
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/wpdsp.h"

#define NAPI_CALL(env, call)                                          \
    do {                                                              \
        if ((call) != napi_ok) {                                      \
            napi_throw_error((env), NULL, "N-API call failed: " #call); \
            return NULL;                                              \
        }                                                             \
    } while (0)

static napi_value throw_wp(napi_env env, int rc)
{
    const char *msg = wp_last_error();
    if (rc == WP_ERR_RANGE) napi_throw_range_error(env, NULL, msg);
    else if (rc == WP_ERR_INVALID) napi_throw_type_error(env, NULL, msg);
    else napi_throw_error(env, NULL, msg);
    return NULL;
}

struct F32View {
    float *data;
    size_t length;
};

static bool get_f32(napi_env env, napi_value v, F32View *out)
{
    bool is = false;
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is) return false;
    napi_typedarray_type type;
    size_t length = 0, byte_offset = 0;
    void *data = NULL;
    napi_value ab;
    if (napi_get_typedarray_info(env, v, &type, &length, &data, &ab, &byte_offset) != napi_ok) return false;
    if (type != napi_float32_array) return false;
    out->data = (float *)data;  // already advanced by byte_offset
    out->length = length;
    return true;
}

static bool get_f64(napi_env env, napi_value v, double **data, size_t *length)
{
    bool is = false;
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is) return false;
    napi_typedarray_type type;
    size_t byte_offset = 0;
    void *d = NULL;
    napi_value ab;
    if (napi_get_typedarray_info(env, v, &type, length, &d, &ab, &byte_offset) != napi_ok) return false;
    if (type != napi_float64_array) return false;
    *data = (double *)d;
    return true;
}

static napi_value new_f32(napi_env env, size_t n, float **data)
{
    napi_value ab, ta;
    void *p = NULL;
    NAPI_CALL(env, napi_create_arraybuffer(env, n * sizeof(float), &p, &ab));
    NAPI_CALL(env, napi_create_typedarray(env, napi_float32_array, n, ab, 0, &ta));
    *data = (float *)p;
    return ta;
}

static bool get_i64(napi_env env, napi_value v, int64_t *out)
{
    return napi_get_value_int64(env, v, out) == napi_ok;
}

static napi_value num(napi_env env, double d)
{
    napi_value v;
    napi_create_double(env, d, &v);
    return v;
}

// ---- handles: JS objects wrapping a native pointer, freed by the GC ----------
static void ir_finalize(napi_env, void *p, void *) { wp_ir_destroy((wp_ir *)p); }
static void reverb_finalize(napi_env, void *p, void *) { wp_reverb_destroy((wp_reverb *)p); }

static napi_value wrap_ptr(napi_env env, void *p, napi_finalize fin)
{
    napi_value obj;
    NAPI_CALL(env, napi_create_object(env, &obj));
    NAPI_CALL(env, napi_wrap(env, obj, p, fin, NULL, NULL));
    return obj;
}

template <class T>
static T *unwrap(napi_env env, napi_value v)
{
    void *p = NULL;
    if (napi_unwrap(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "expected a wpdsp handle");
        return NULL;
    }
    return (T *)p;
}

#define ARGS(n)                                                          \
    size_t argc = (n);                                                   \
    napi_value argv[(n)];                                                \
    NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL)); \
    if (argc < (n)) {                                                    \
        napi_throw_type_error(env, NULL, "too few arguments");           \
        return NULL;                                                     \
    }

// init() -> number of GPUs
static napi_value js_init(napi_env env, napi_callback_info)
{
    const int n = wp_init();
    if (n < 0) return throw_wp(env, n);
    return num(env, n);
}

// fftLength(L, M) -> N   (lib/convolve.js:554-555)
static napi_value js_fft_length(napi_env env, napi_callback_info info)
{
    ARGS(2);
    int64_t L, M;
    if (!get_i64(env, argv[0], &L) || !get_i64(env, argv[1], &M)) return throw_wp(env, WP_ERR_INVALID);
    const int32_t n = wp_fft_length(L, M);
    if (n < 0) {
        napi_throw_error(env, NULL, "FFT size must be a power of two and bigger than 1");
        return NULL;
    }
    return num(env, n);
}

// irCreate([Float32Array, ...], L) -> handle   (lib/convolve.js:553-562)
static napi_value js_ir_create(napi_env env, napi_callback_info info)
{
    ARGS(2);
    uint32_t nch = 0;
    NAPI_CALL(env, napi_get_array_length(env, argv[0], &nch));
    int64_t L;
    if (!get_i64(env, argv[1], &L) || nch < 1) return throw_wp(env, WP_ERR_INVALID);
    std::vector<const float *> ptrs(nch);
    size_t M = 0;
    for (uint32_t c = 0; c < nch; c++) {
        napi_value e;
        NAPI_CALL(env, napi_get_element(env, argv[0], c, &e));
        F32View v;
        if (!get_f32(env, e, &v) || (c > 0 && v.length != M)) {
            napi_throw_type_error(env, NULL, "IR channels must be Float32Arrays of equal length");
            return NULL;
        }
        M = v.length;
        ptrs[c] = v.data;
    }
    wp_ir *ir = wp_ir_create(ptrs.data(), (int)nch, (int64_t)M, L);
    if (!ir) return throw_wp(env, WP_ERR_HIP);
    return wrap_ptr(env, ir, ir_finalize);
}

// irInfo(handle, which)
static napi_value js_ir_info(napi_env env, napi_callback_info info)
{
    ARGS(2);
    wp_ir *ir = unwrap<wp_ir>(env, argv[0]);
    if (!ir) return NULL;
    int64_t which;
    if (!get_i64(env, argv[1], &which)) return throw_wp(env, WP_ERR_INVALID);
    return num(env, (double)wp_ir_info(ir, (int)which));
}

// convolve(ir, ch, input) -> Float32Array(L+M)   (go, lib/convolve.js:570-583)
static napi_value js_convolve(napi_env env, napi_callback_info info)
{
    ARGS(3);
    wp_ir *ir = unwrap<wp_ir>(env, argv[0]);
    if (!ir) return NULL;
    int64_t ch;
    F32View in;
    if (!get_i64(env, argv[1], &ch) || !get_f32(env, argv[2], &in)) {
        napi_throw_type_error(env, NULL, "convolve(ir, ch, Float32Array)");
        return NULL;
    }
    float *out = NULL;
    napi_value ta = new_f32(env, (size_t)wp_ir_info(ir, 5), &out);
    if (!ta) return NULL;
    const int rc = wp_convolve(ir, (int)ch, in.data, (int64_t)in.length, out);
    if (rc) return throw_wp(env, rc);
    return ta;
}

// reverbCreate(ir, nch, offset) -> handle
static napi_value js_reverb_create(napi_env env, napi_callback_info info)
{
    ARGS(3);
    wp_ir *ir = unwrap<wp_ir>(env, argv[0]);
    if (!ir) return NULL;
    int64_t nch, offset;
    if (!get_i64(env, argv[1], &nch) || !get_i64(env, argv[2], &offset) || nch < 1 || nch > 2) return throw_wp(env, WP_ERR_INVALID);
    const int32_t job_ch[2] = {0, 1};
    wp_reverb *r = wp_reverb_create(ir, (int)nch, job_ch, offset);
    if (!r) return throw_wp(env, WP_ERR_HIP);
    // the reverb borrows the IR: keep the IR object alive as a property of the handle
    napi_value obj = wrap_ptr(env, r, reverb_finalize);
    if (obj) {
        napi_set_named_property(env, obj, "ir", argv[0]);
        napi_set_named_property(env, obj, "jobs", num(env, (double)nch));
    }
    return obj;
}

// number of channel-jobs a reverb handle was created with (every per-job array must match it)
static size_t reverb_jobs(napi_env env, napi_value handle)
{
    napi_value v;
    int64_t n = 0;
    if (napi_get_named_property(env, handle, "jobs", &v) != napi_ok || !get_i64(env, v, &n)) return 0;
    return (size_t)n;
}

static bool collect_channels(napi_env env, napi_value arr, std::vector<F32View> *out)
{
    uint32_t n = 0;
    if (napi_get_array_length(env, arr, &n) != napi_ok) return false;
    out->resize(n);
    for (uint32_t c = 0; c < n; c++) {
        napi_value e;
        if (napi_get_element(env, arr, c, &e) != napi_ok || !get_f32(env, e, &(*out)[c])) return false;
    }
    return true;
}

// reverbBlock(handle, [Float32Array per channel]) ->
//   { curr: [Float32Array(L+M)...], retOff, retLen, prevOff, prevLen }
static napi_value js_reverb_block(napi_env env, napi_callback_info info)
{
    ARGS(2);
    wp_reverb *r = unwrap<wp_reverb>(env, argv[0]);
    if (!r) return NULL;
    std::vector<F32View> in;
    if (!collect_channels(env, argv[1], &in) || in.empty() || in.size() != reverb_jobs(env, argv[0])) {
        napi_throw_type_error(env, NULL, "reverbBlock(handle, [one Float32Array per channel])");
        return NULL;
    }
    napi_value irv;
    NAPI_CALL(env, napi_get_named_property(env, argv[0], "ir", &irv));
    wp_ir *ir = unwrap<wp_ir>(env, irv);
    if (!ir) return NULL;
    const size_t n = (size_t)wp_ir_info(ir, 5);
    std::vector<const float *> ip(in.size());
    std::vector<float *> op(in.size());
    napi_value curr;
    NAPI_CALL(env, napi_create_array_with_length(env, in.size(), &curr));
    for (size_t c = 0; c < in.size(); c++) {
        if (in[c].length != in[0].length) {
            napi_throw_type_error(env, NULL, "channels must have equal length");
            return NULL;
        }
        ip[c] = in[c].data;
        napi_value ta = new_f32(env, n, &op[c]);
        if (!ta) return NULL;
        NAPI_CALL(env, napi_set_element(env, curr, (uint32_t)c, ta));
    }
    wp_reverb_view v;
    const int rc = wp_reverb_block(r, ip.data(), (int64_t)in[0].length, op.data(), &v);
    if (rc) return throw_wp(env, rc);
    napi_value res;
    NAPI_CALL(env, napi_create_object(env, &res));
    napi_set_named_property(env, res, "curr", curr);
    napi_set_named_property(env, res, "retOff", num(env, (double)v.ret_off));
    napi_set_named_property(env, res, "retLen", num(env, (double)v.ret_len));
    napi_set_named_property(env, res, "prevOff", num(env, (double)v.prev_off));
    napi_set_named_property(env, res, "prevLen", num(env, (double)v.prev_len));
    return res;
}

// reverbSetTail(handle, [Float32Array per channel])
static napi_value js_reverb_set_tail(napi_env env, napi_callback_info info)
{
    ARGS(2);
    wp_reverb *r = unwrap<wp_reverb>(env, argv[0]);
    if (!r) return NULL;
    std::vector<F32View> pv;
    if (!collect_channels(env, argv[1], &pv) || pv.empty() || pv.size() != reverb_jobs(env, argv[0])) {
        napi_throw_type_error(env, NULL, "reverbSetTail(handle, [one Float32Array per channel])");
        return NULL;
    }
    std::vector<const float *> pp(pv.size());
    for (size_t c = 0; c < pv.size(); c++) {
        if (pv[c].length < pv[0].length) {
            napi_throw_type_error(env, NULL, "reverbSetTail: tails must have equal length");
            return NULL;
        }
        pp[c] = pv[c].data;
    }
    const int rc = wp_reverb_set_tail(r, pp.data(), (int64_t)pv[0].length);
    if (rc) return throw_wp(env, rc);
    return NULL;
}

// mix(target: [Float32Array...], sources: [{ch:[Float32Array...], rl, sl, vol, volVec?, o}])
static napi_value js_mix(napi_env env, napi_callback_info info)
{
    ARGS(2);
    std::vector<F32View> tgt;
    if (!collect_channels(env, argv[0], &tgt) || tgt.empty()) {
        napi_throw_type_error(env, NULL, "mix: target must be an array of Float32Arrays");
        return NULL;
    }
    uint32_t nsrc = 0;
    NAPI_CALL(env, napi_get_array_length(env, argv[1], &nsrc));
    std::vector<wp_src> srcs(nsrc);
    for (uint32_t i = 0; i < nsrc; i++) {
        napi_value s, v;
        NAPI_CALL(env, napi_get_element(env, argv[1], i, &s));
        wp_src &d = srcs[i];
        memset(&d, 0, sizeof(d));
        NAPI_CALL(env, napi_get_named_property(env, s, "ch", &v));
        std::vector<F32View> ch;
        if (!collect_channels(env, v, &ch)) {
            napi_throw_type_error(env, NULL, "mix: source channels must be Float32Arrays");
            return NULL;
        }
        d.nch = (int32_t)ch.size();
        for (size_t c = 0; c < ch.size() && c < 2; c++) d.ch[c] = ch[c].data;
        int64_t t;
        double f;
        NAPI_CALL(env, napi_get_named_property(env, s, "rl", &v));
        if (!get_i64(env, v, &t)) return throw_wp(env, WP_ERR_INVALID);
        d.rl = t;
        // rl frames are read from every channel: a shorter second channel (NaN past its end in
        // the reference) is refused rather than read out of bounds
        for (size_t c = 0; c < ch.size() && c < 2; c++)
            if (t < 0 || (int64_t)ch[c].length < t) {
                napi_throw_type_error(env, NULL, "mix: source channels must all hold source[0].length frames");
                return NULL;
            }
        NAPI_CALL(env, napi_get_named_property(env, s, "sl", &v));
        if (!get_i64(env, v, &t)) return throw_wp(env, WP_ERR_INVALID);
        d.sl = t;
        NAPI_CALL(env, napi_get_named_property(env, s, "o", &v));
        if (!get_i64(env, v, &t)) return throw_wp(env, WP_ERR_INVALID);
        d.o = t;
        NAPI_CALL(env, napi_get_named_property(env, s, "vol", &v));
        if (napi_get_value_double(env, v, &f) != napi_ok) return throw_wp(env, WP_ERR_INVALID);
        d.vol = f;
        bool has = false;
        NAPI_CALL(env, napi_has_named_property(env, s, "volVec", &has));
        if (has) {
            NAPI_CALL(env, napi_get_named_property(env, s, "volVec", &v));
            napi_valuetype vt;
            NAPI_CALL(env, napi_typeof(env, v, &vt));
            if (vt != napi_undefined && vt != napi_null) {
                double *vec = NULL;
                size_t vl = 0;
                if (!get_f64(env, v, &vec, &vl) || vl < tgt[0].length) {
                    napi_throw_type_error(env, NULL, "mix: volVec must be a Float64Array of the target length");
                    return NULL;
                }
                d.vol_vec = vec;
            }
        }
    }
    std::vector<float *> tp(tgt.size());
    for (size_t c = 0; c < tgt.size(); c++) {
        if (tgt[c].length < tgt[0].length) {
            napi_throw_type_error(env, NULL, "mix: target channels must all hold target[0].length frames");
            return NULL;
        }
        tp[c] = tgt[c].data;
    }
    const int rc = wp_mix(tp.data(), (int)tgt.size(), (int64_t)tgt[0].length, srcs.data(), (int64_t)nsrc);
    if (rc) return throw_wp(env, rc);
    return argv[0];
}

// mixSl(rl, length) -> (rl*length)|0
static napi_value js_mix_sl(napi_env env, napi_callback_info info)
{
    ARGS(2);
    int64_t rl;
    double len;
    if (!get_i64(env, argv[0], &rl) || napi_get_value_double(env, argv[1], &len) != napi_ok) return throw_wp(env, WP_ERR_INVALID);
    return num(env, wp_mix_sl(rl, len));
}


// ---- device-resident buffers ----------------------------------------------------
// A device channel is described from JavaScript as { h: <buffer handle>, ch, off, length }
// (dsp_amd/js/src/device-buffer.js); it resolves to a device pointer and a frame count.
// V8 is told that device memory hangs off each handle, so a loop that drops a buffer per
// block triggers garbage collection before HBM fills up.  It is told an eighth of the
// size: the collector weighs external bytes as if they were host memory (a full
// collection per ~64 MB), which for 1.5 MB results meant one mark-sweep every 45 bars,
// about 20 us per bar; at an eighth some 500 MB of dropped buffers wait for a collection,
// little against the device's memory.
static int64_t reported_bytes(const wp_buf *b) { return wp_buf_info(b, 4) >> 3; }
static void buf_finalize(napi_env env, void *p, void *)
{
    int64_t adjusted = 0;
    napi_adjust_external_memory(env, -reported_bytes((wp_buf *)p), &adjusted);
    wp_buf_release((wp_buf *)p);
}

static napi_value wrap_buf(napi_env env, wp_buf *b)
{
    napi_value obj = wrap_ptr(env, b, buf_finalize);
    if (!obj) {
        wp_buf_release(b);
        return NULL;
    }
    int64_t adjusted = 0;
    napi_adjust_external_memory(env, reported_bytes(b), &adjusted);
    return obj;
}

struct DevChan {
    wp_buf *buf;
    int ch;
    int64_t off, len;
    float *ptr;
};

static bool get_dev_chan(napi_env env, napi_value v, DevChan *out)
{
    napi_valuetype vt;
    if (napi_typeof(env, v, &vt) != napi_ok || vt != napi_object) return false;
    napi_value h, f;
    int64_t t;
    if (napi_get_named_property(env, v, "h", &h) != napi_ok) return false;
    void *p = NULL;
    if (napi_unwrap(env, h, &p) != napi_ok || !p) return false;
    out->buf = (wp_buf *)p;
    if (napi_get_named_property(env, v, "ch", &f) != napi_ok || !get_i64(env, f, &t)) return false;
    out->ch = (int)t;
    if (napi_get_named_property(env, v, "off", &f) != napi_ok || !get_i64(env, f, &out->off)) return false;
    if (napi_get_named_property(env, v, "length", &f) != napi_ok || !get_i64(env, f, &out->len)) return false;
    if (out->ch < 0 || out->ch >= wp_buf_info(out->buf, 0)) return false;
    if (out->off < 0 || out->len < 0 || out->off + out->len > wp_buf_info(out->buf, 1)) return false;
    out->ptr = wp_buf_channel(out->buf, out->ch) + out->off;
    return true;
}

static bool collect_dev_chans(napi_env env, napi_value arr, std::vector<DevChan> *out)
{
    uint32_t n = 0;
    if (napi_get_array_length(env, arr, &n) != napi_ok) return false;
    out->resize(n);
    for (uint32_t c = 0; c < n; c++) {
        napi_value e;
        if (napi_get_element(env, arr, c, &e) != napi_ok || !get_dev_chan(env, e, &(*out)[c])) return false;
    }
    return true;
}

// bufCreate(nch, frames) -> handle
static napi_value js_buf_create(napi_env env, napi_callback_info info)
{
    ARGS(2);
    int64_t nch, frames;
    if (!get_i64(env, argv[0], &nch) || !get_i64(env, argv[1], &frames)) return throw_wp(env, WP_ERR_INVALID);
    wp_buf *b = wp_buf_create((int)nch, frames);
    if (!b) return throw_wp(env, WP_ERR_HIP);
    return wrap_buf(env, b);
}

// bufOpen(id) -> handle (another reference to a buffer created by any thread)
static napi_value js_buf_open(napi_env env, napi_callback_info info)
{
    ARGS(1);
    int64_t id;
    if (!get_i64(env, argv[0], &id) || id <= 0) return throw_wp(env, WP_ERR_INVALID);
    wp_buf *b = wp_buf_open((uint64_t)id);
    if (!b) return throw_wp(env, WP_ERR_HIP);
    return wrap_buf(env, b);
}

// bufInfo(handle, which) ; which == -1: the process-wide id
static napi_value js_buf_info(napi_env env, napi_callback_info info)
{
    ARGS(2);
    wp_buf *b = unwrap<wp_buf>(env, argv[0]);
    if (!b) return NULL;
    int64_t which;
    if (!get_i64(env, argv[1], &which)) return throw_wp(env, WP_ERR_INVALID);
    if (which == -1) return num(env, (double)wp_buf_id(b));
    return num(env, (double)wp_buf_info(b, (int)which));
}

// bufWrite(handle, ch, dstOff, Float32Array)
static napi_value js_buf_write(napi_env env, napi_callback_info info)
{
    ARGS(4);
    wp_buf *b = unwrap<wp_buf>(env, argv[0]);
    if (!b) return NULL;
    int64_t ch, off;
    F32View src;
    if (!get_i64(env, argv[1], &ch) || !get_i64(env, argv[2], &off) || !get_f32(env, argv[3], &src)) {
        napi_throw_type_error(env, NULL, "bufWrite(handle, ch, offset, Float32Array)");
        return NULL;
    }
    const int rc = wp_buf_write(b, (int)ch, off, src.data, (int64_t)src.length);
    if (rc) return throw_wp(env, rc);
    return NULL;
}

// bufRead(handle, ch, off, n) -> Float32Array(n)
static napi_value js_buf_read(napi_env env, napi_callback_info info)
{
    ARGS(4);
    wp_buf *b = unwrap<wp_buf>(env, argv[0]);
    if (!b) return NULL;
    int64_t ch, off, n;
    if (!get_i64(env, argv[1], &ch) || !get_i64(env, argv[2], &off) || !get_i64(env, argv[3], &n) || n < 0)
        return throw_wp(env, WP_ERR_INVALID);
    float *out = NULL;
    napi_value ta = new_f32(env, (size_t)n, &out);
    if (!ta) return NULL;
    const int rc = wp_buf_read(b, (int)ch, off, out, n);
    if (rc) return throw_wp(env, rc);
    return ta;
}

// bufCopy(dstChan, srcChan, dstOff): dst.set(src, dstOff) between device channels
static napi_value js_buf_copy(napi_env env, napi_callback_info info)
{
    ARGS(3);
    DevChan d, s;
    int64_t doff;
    if (!get_dev_chan(env, argv[0], &d) || !get_dev_chan(env, argv[1], &s) || !get_i64(env, argv[2], &doff)) {
        napi_throw_type_error(env, NULL, "bufCopy(dstChannel, srcChannel, offset)");
        return NULL;
    }
    if (doff < 0 || doff + s.len > d.len) {
        napi_throw_range_error(env, NULL, "offset is out of bounds");
        return NULL;
    }
    const int rc = wp_buf_copy(d.buf, d.ch, d.off + doff, s.buf, s.ch, s.off, s.len, NULL);
    if (rc) return throw_wp(env, rc);
    return NULL;
}

// bufFill(chan, value, start, end)
static napi_value js_buf_fill(napi_env env, napi_callback_info info)
{
    ARGS(4);
    DevChan d;
    double value;
    int64_t a, b;
    if (!get_dev_chan(env, argv[0], &d) || napi_get_value_double(env, argv[1], &value) != napi_ok ||
        !get_i64(env, argv[2], &a) || !get_i64(env, argv[3], &b) || a < 0 || b > d.len) {
        napi_throw_type_error(env, NULL, "bufFill(channel, value, start, end)");
        return NULL;
    }
    if (b > a) {
        const int rc = wp_buf_fill(d.buf, d.ch, d.off + a, b - a, (float)value, NULL);
        if (rc) return throw_wp(env, rc);
    }
    return NULL;
}

// convolveDev(ir, irChannel, inChan, outChan): go() with input and result on the device
static napi_value js_convolve_dev(napi_env env, napi_callback_info info)
{
    ARGS(4);
    wp_ir *ir = unwrap<wp_ir>(env, argv[0]);
    if (!ir) return NULL;
    int64_t ch;
    DevChan in, out;
    if (!get_i64(env, argv[1], &ch) || !get_dev_chan(env, argv[2], &in) || !get_dev_chan(env, argv[3], &out) ||
        out.len < wp_ir_info(ir, 5)) {
        napi_throw_type_error(env, NULL, "convolveDev(ir, ch, inChannel, outChannel[L+M])");
        return NULL;
    }
    const int rc = wp_convolve_dev(ir, (int)ch, in.ptr, in.len, out.ptr, NULL);
    if (rc) return throw_wp(env, rc);
    return NULL;
}

// equal spacing of job pointers: job j at p[0] + j*stride
static bool job_stride(const std::vector<DevChan> &c, int64_t *stride)
{
    *stride = c.size() > 1 ? (int64_t)(c[1].ptr - c[0].ptr) : 0;
    for (size_t j = 1; j < c.size(); j++) {
        if (c[j].len != c[0].len) return false;
        if ((int64_t)(c[j].ptr - c[0].ptr) != *stride * (int64_t)j) return false;
    }
    return true;
}

// reverbBlockTo(state, [inChan per job], [outChan per job]) -> { retOff, retLen, prevOff, prevLen }
// The result (tail already added) lands in the out channels, each L+M frames; the
// state keeps them referenced because the carried tail is a view into them.
static napi_value js_reverb_block_to(napi_env env, napi_callback_info info)
{
    ARGS(3);
    wp_reverb *r = unwrap<wp_reverb>(env, argv[0]);
    if (!r) return NULL;
    std::vector<DevChan> in, out;
    int64_t is = 0, os = 0;
    if (!collect_dev_chans(env, argv[1], &in) || !collect_dev_chans(env, argv[2], &out) || in.empty() ||
        in.size() != out.size() || in.size() != reverb_jobs(env, argv[0]) || !job_stride(in, &is) || !job_stride(out, &os)) {
        napi_throw_type_error(env, NULL, "reverbBlockTo(state, [in channels], [out channels]): channels of one buffer, equal lengths");
        return NULL;
    }
    napi_value irv;
    NAPI_CALL(env, napi_get_named_property(env, argv[0], "ir", &irv));
    wp_ir *ir = unwrap<wp_ir>(env, irv);
    if (!ir) return NULL;
    if (out[0].len < wp_ir_info(ir, 5)) {
        napi_throw_range_error(env, NULL, "reverbBlockTo: result channels need L+M frames");
        return NULL;
    }
    wp_reverb_view v;
    const int rc = wp_reverb_block_to(r, in[0].ptr, is, in[0].len, out[0].ptr, os, &v, NULL);
    if (rc) return throw_wp(env, rc);
    napi_set_named_property(env, argv[0], "tailOwner", argv[2]);
    napi_value res;
    NAPI_CALL(env, napi_create_object(env, &res));
    napi_set_named_property(env, res, "retOff", num(env, (double)v.ret_off));
    napi_set_named_property(env, res, "retLen", num(env, (double)v.ret_len));
    napi_set_named_property(env, res, "prevOff", num(env, (double)v.prev_off));
    napi_set_named_property(env, res, "prevLen", num(env, (double)v.prev_len));
    return res;
}

// reverbSetTailDev(state, [chan per job]): carried tail taken from device channels, no copy
static napi_value js_reverb_set_tail_dev(napi_env env, napi_callback_info info)
{
    ARGS(2);
    wp_reverb *r = unwrap<wp_reverb>(env, argv[0]);
    if (!r) return NULL;
    std::vector<DevChan> pv;
    int64_t ps = 0;
    if (!collect_dev_chans(env, argv[1], &pv) || pv.empty() || pv.size() != reverb_jobs(env, argv[0]) || !job_stride(pv, &ps)) {
        napi_throw_type_error(env, NULL, "reverbSetTailDev(state, [channels of one buffer])");
        return NULL;
    }
    const int rc = wp_reverb_set_tail_dev(r, pv[0].ptr, ps, pv[0].len);
    if (rc) return throw_wp(env, rc);
    napi_set_named_property(env, argv[0], "tailOwner", argv[1]);
    return NULL;
}

// mixDev(target: [chan...], sources: [{ch:[chan...], sl, vol, volVec?, o}]) -- everything on the device
static napi_value js_mix_dev(napi_env env, napi_callback_info info)
{
    ARGS(2);
    std::vector<DevChan> tgt;
    if (!collect_dev_chans(env, argv[0], &tgt) || tgt.empty()) {
        napi_throw_type_error(env, NULL, "mixDev: target must be an array of device channels");
        return NULL;
    }
    const int64_t tl = tgt[0].len;
    uint32_t nsrc = 0;
    NAPI_CALL(env, napi_get_array_length(env, argv[1], &nsrc));
    std::vector<wp_src> srcs(nsrc);
    std::vector<wp_buf *> temps;  // gain vectors staged on the device for this call
    auto drop_temps = [&]() {
        for (wp_buf *t : temps) wp_buf_release(t);
    };
    for (uint32_t i = 0; i < nsrc; i++) {
        napi_value s, v;
        NAPI_CALL(env, napi_get_element(env, argv[1], i, &s));
        wp_src &d = srcs[i];
        memset(&d, 0, sizeof(d));
        std::vector<DevChan> ch;
        if (napi_get_named_property(env, s, "ch", &v) != napi_ok || !collect_dev_chans(env, v, &ch)) {
            drop_temps();
            napi_throw_type_error(env, NULL, "mixDev: source channels must be device channels");
            return NULL;
        }
        d.nch = (int32_t)ch.size();
        for (size_t c = 0; c < ch.size() && c < 2; c++) d.ch[c] = ch[c].ptr;
        d.rl = ch.empty() ? 0 : ch[0].len;
        for (size_t c = 1; c < ch.size() && c < 2; c++)
            if (ch[c].len < d.rl) {
                drop_temps();
                napi_throw_type_error(env, NULL, "mixDev: source channels must all hold source[0].length frames");
                return NULL;
            }
        int64_t t;
        double f;
        if (napi_get_named_property(env, s, "sl", &v) != napi_ok || !get_i64(env, v, &t)) { drop_temps(); return throw_wp(env, WP_ERR_INVALID); }
        d.sl = t;
        if (napi_get_named_property(env, s, "o", &v) != napi_ok || !get_i64(env, v, &t)) { drop_temps(); return throw_wp(env, WP_ERR_INVALID); }
        d.o = t;
        if (napi_get_named_property(env, s, "vol", &v) != napi_ok || napi_get_value_double(env, v, &f) != napi_ok) { drop_temps(); return throw_wp(env, WP_ERR_INVALID); }
        d.vol = f;
        bool has = false;
        napi_has_named_property(env, s, "volVec", &has);
        if (has) {
            napi_get_named_property(env, s, "volVec", &v);
            napi_valuetype vt;
            napi_typeof(env, v, &vt);
            if (vt != napi_undefined && vt != napi_null) {
                double *vec = NULL;
                size_t vl = 0;
                if (!get_f64(env, v, &vec, &vl) || (int64_t)vl < tl) {
                    drop_temps();
                    napi_throw_type_error(env, NULL, "mixDev: volVec must be a Float64Array of the target length");
                    return NULL;
                }
                // tl doubles travel as 2*tl floats' worth of bytes
                wp_buf *t2 = wp_buf_create(1, 2 * tl);
                if (!t2 || wp_buf_write(t2, 0, 0, (const float *)vec, 2 * tl)) {
                    if (t2) wp_buf_release(t2);
                    drop_temps();
                    return throw_wp(env, WP_ERR_HIP);
                }
                temps.push_back(t2);
                d.vol_vec = (const double *)wp_buf_channel(t2, 0);
            }
        }
    }
    float *tp[2] = {tgt[0].ptr, tgt.size() > 1 ? tgt[1].ptr : NULL};
    if (tgt.size() == 2 && tgt[1].len < tl) {
        drop_temps();
        napi_throw_type_error(env, NULL, "mixDev: target channels must all hold target[0].length frames");
        return NULL;
    }
    int rc = WP_OK;
    if (tgt.size() <= 2) rc = wp_mix_dev(tp, (int)tgt.size(), tl, srcs.data(), (int64_t)nsrc, 0, NULL);
    drop_temps();  // recycled memory is only reused in stream order, after the mix kernel
    if (rc) return throw_wp(env, rc);
    return argv[0];
}

static napi_value js_ring_pos(napi_env env, napi_callback_info info)
{
    ARGS(3);
    int64_t a, b, c;
    if (!get_i64(env, argv[0], &a) || !get_i64(env, argv[1], &b) || !get_i64(env, argv[2], &c)) return throw_wp(env, WP_ERR_INVALID);
    return num(env, (double)wp_ring_pos(a, b, c));
}

static napi_value js_beat_rate(napi_env env, napi_callback_info info)
{
    ARGS(2);
    double sr, bpm;
    if (napi_get_value_double(env, argv[0], &sr) != napi_ok || napi_get_value_double(env, argv[1], &bpm) != napi_ok)
        return throw_wp(env, WP_ERR_INVALID);
    return num(env, (double)wp_beat_rate(sr, bpm));
}

static napi_value js_bpm_from_beat_rate(napi_env env, napi_callback_info info)
{
    ARGS(2);
    double sr;
    int64_t br;
    if (napi_get_value_double(env, argv[0], &sr) != napi_ok || !get_i64(env, argv[1], &br)) return throw_wp(env, WP_ERR_INVALID);
    return num(env, wp_bpm_from_beat_rate(sr, br));
}

// directConvolution(input, kernel) -> Float32Array(n+m-1), bit-identical to lib/convolve.js:541-551
static napi_value js_direct_convolution(napi_env env, napi_callback_info info)
{
    ARGS(2);
    F32View x, h;
    if (!get_f32(env, argv[0], &x) || !get_f32(env, argv[1], &h)) {
        napi_throw_type_error(env, NULL, "directConvolution(Float32Array, Float32Array)");
        return NULL;
    }
    const size_t len = x.length + h.length > 0 ? x.length + h.length - 1 : 0;
    float *out = NULL;
    napi_value ta = new_f32(env, len, &out);
    if (!ta) return NULL;
    const int rc = wp_direct_convolution(x.data, (int64_t)x.length, h.data, (int64_t)h.length, out);
    if (rc) return throw_wp(env, rc);
    return ta;
}

// kernelSpectrum(kernel, L, out): the reference-layout spectrum (2N floats) into `out`
static napi_value js_kernel_spectrum(napi_env env, napi_callback_info info)
{
    ARGS(3);
    F32View h, out;
    int64_t L;
    if (!get_f32(env, argv[0], &h) || !get_i64(env, argv[1], &L) || !get_f32(env, argv[2], &out)) {
        napi_throw_type_error(env, NULL, "kernelSpectrum(Float32Array, length, Float32Array)");
        return NULL;
    }
    const int32_t N = wp_fft_length(L, (int64_t)h.length);
    if (N < 0) return throw_wp(env, N);
    if (out.length < 2 * (size_t)N) {
        napi_throw_range_error(env, NULL, "kernelSpectrum: the output holds fewer than 2N floats");
        return NULL;
    }
    const int rc = wp_kernel_spectrum(h.data, (int64_t)h.length, L, out.data);
    if (rc) return throw_wp(env, rc);
    return argv[2];
}

// kernelFromSpectrum(spectrum /* 2N */, M) -> Float32Array(M)
static napi_value js_kernel_from_spectrum(napi_env env, napi_callback_info info)
{
    ARGS(2);
    F32View spec;
    int64_t M;
    if (!get_f32(env, argv[0], &spec) || !get_i64(env, argv[1], &M) || M < 1 || spec.length % 2) {
        napi_throw_type_error(env, NULL, "kernelFromSpectrum(Float32Array(2N), M)");
        return NULL;
    }
    float *out = NULL;
    napi_value ta = new_f32(env, (size_t)M, &out);
    if (!ta) return NULL;
    const int rc = wp_kernel_from_spectrum(spec.data, (int64_t)(spec.length / 2), M, out);
    if (rc) return throw_wp(env, rc);
    return ta;
}

// ---- multi-GPU: one process per GPU, the partial master buses summed by RCCL ------
static void comm_finalize(napi_env, void *p, void *) { wp_comm_destroy((wp_comm *)p); }

// commUniqueId() -> Uint8Array(128), made by rank 0 and handed to the other ranks by the host
static napi_value js_comm_unique_id(napi_env env, napi_callback_info)
{
    napi_value ab, ta;
    void *p = NULL;
    NAPI_CALL(env, napi_create_arraybuffer(env, WP_COMM_ID_BYTES, &p, &ab));
    const int rc = wp_comm_unique_id((unsigned char *)p);
    if (rc) return throw_wp(env, rc);
    NAPI_CALL(env, napi_create_typedarray(env, napi_uint8_array, WP_COMM_ID_BYTES, ab, 0, &ta));
    return ta;
}

// commCreate(idUint8Array, nranks, rank) -> handle (returns when every rank has joined)
static napi_value js_comm_create(napi_env env, napi_callback_info info)
{
    ARGS(3);
    bool is = false;
    napi_typedarray_type type;
    size_t length = 0, byte_offset = 0;
    void *data = NULL;
    napi_value ab;
    int64_t nranks, rank;
    if (napi_is_typedarray(env, argv[0], &is) != napi_ok || !is ||
        napi_get_typedarray_info(env, argv[0], &type, &length, &data, &ab, &byte_offset) != napi_ok ||
        type != napi_uint8_array || length != WP_COMM_ID_BYTES || !get_i64(env, argv[1], &nranks) ||
        !get_i64(env, argv[2], &rank)) {
        napi_throw_type_error(env, NULL, "commCreate(Uint8Array(128) id, nranks, rank)");
        return NULL;
    }
    wp_comm *c = wp_comm_create((const unsigned char *)data, (int)nranks, (int)rank);
    if (!c) return throw_wp(env, WP_ERR_HIP);
    return wrap_ptr(env, c, comm_finalize);
}

// busReduce(comm, bufferHandle, root, all): every channel of a device buffer (one message) summed over
// the ranks, onto `root` in place, or onto every rank when `all` is true.  Returns when the result is there.
static napi_value js_bus_reduce(napi_env env, napi_callback_info info)
{
    ARGS(4);
    wp_comm *c = unwrap<wp_comm>(env, argv[0]);
    if (!c) return NULL;
    wp_buf *b = unwrap<wp_buf>(env, argv[1]);
    if (!b) return NULL;
    int64_t root;
    bool all = false;
    if (!get_i64(env, argv[2], &root) || napi_get_value_bool(env, argv[3], &all) != napi_ok) return throw_wp(env, WP_ERR_INVALID);
    float *p = wp_buf_channel(b, 0);
    const int64_t count = wp_buf_info(b, 0) * wp_buf_info(b, 2);  // channels x floats between channels (padding is zero)
    int rc = all ? wp_bus_allreduce(c, p, p, count, NULL) : wp_bus_reduce(c, p, p, count, (int)root, NULL);
    if (rc) return throw_wp(env, rc);
    float probe;  // a blocking read on the default stream: the reduce has completed when it returns
    rc = wp_buf_info(b, 1) > 0 ? wp_buf_read(b, 0, 0, &probe, 1) : 0;
    if (rc) return throw_wp(env, rc);
    return NULL;
}

NAPI_MODULE_INIT()
{
    const napi_property_descriptor props[] = {
        {"init", NULL, js_init, NULL, NULL, NULL, napi_enumerable, NULL},
        {"fftLength", NULL, js_fft_length, NULL, NULL, NULL, napi_enumerable, NULL},
        {"irCreate", NULL, js_ir_create, NULL, NULL, NULL, napi_enumerable, NULL},
        {"irInfo", NULL, js_ir_info, NULL, NULL, NULL, napi_enumerable, NULL},
        {"convolve", NULL, js_convolve, NULL, NULL, NULL, napi_enumerable, NULL},
        {"reverbCreate", NULL, js_reverb_create, NULL, NULL, NULL, napi_enumerable, NULL},
        {"reverbBlock", NULL, js_reverb_block, NULL, NULL, NULL, napi_enumerable, NULL},
        {"reverbSetTail", NULL, js_reverb_set_tail, NULL, NULL, NULL, napi_enumerable, NULL},
        {"mix", NULL, js_mix, NULL, NULL, NULL, napi_enumerable, NULL},
        {"mixSl", NULL, js_mix_sl, NULL, NULL, NULL, napi_enumerable, NULL},
        {"bufCreate", NULL, js_buf_create, NULL, NULL, NULL, napi_enumerable, NULL},
        {"bufOpen", NULL, js_buf_open, NULL, NULL, NULL, napi_enumerable, NULL},
        {"bufInfo", NULL, js_buf_info, NULL, NULL, NULL, napi_enumerable, NULL},
        {"bufWrite", NULL, js_buf_write, NULL, NULL, NULL, napi_enumerable, NULL},
        {"bufRead", NULL, js_buf_read, NULL, NULL, NULL, napi_enumerable, NULL},
        {"bufCopy", NULL, js_buf_copy, NULL, NULL, NULL, napi_enumerable, NULL},
        {"bufFill", NULL, js_buf_fill, NULL, NULL, NULL, napi_enumerable, NULL},
        {"convolveDev", NULL, js_convolve_dev, NULL, NULL, NULL, napi_enumerable, NULL},
        {"reverbBlockTo", NULL, js_reverb_block_to, NULL, NULL, NULL, napi_enumerable, NULL},
        {"reverbSetTailDev", NULL, js_reverb_set_tail_dev, NULL, NULL, NULL, napi_enumerable, NULL},
        {"mixDev", NULL, js_mix_dev, NULL, NULL, NULL, napi_enumerable, NULL},
        {"ringPos", NULL, js_ring_pos, NULL, NULL, NULL, napi_enumerable, NULL},
        {"beatRate", NULL, js_beat_rate, NULL, NULL, NULL, napi_enumerable, NULL},
        {"bpmFromBeatRate", NULL, js_bpm_from_beat_rate, NULL, NULL, NULL, napi_enumerable, NULL},
        {"directConvolution", NULL, js_direct_convolution, NULL, NULL, NULL, napi_enumerable, NULL},
        {"kernelSpectrum", NULL, js_kernel_spectrum, NULL, NULL, NULL, napi_enumerable, NULL},
        {"kernelFromSpectrum", NULL, js_kernel_from_spectrum, NULL, NULL, NULL, napi_enumerable, NULL},
        {"commUniqueId", NULL, js_comm_unique_id, NULL, NULL, NULL, napi_enumerable, NULL},
        {"commCreate", NULL, js_comm_create, NULL, NULL, NULL, napi_enumerable, NULL},
        {"busReduce", NULL, js_bus_reduce, NULL, NULL, NULL, napi_enumerable, NULL},
    };
    if (napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props) != napi_ok) return NULL;
    return exports;
}
