// Host-side runtime bits of libapex_mi355.so: error text, launch checks and the HIP-event
// kernel timer behind apexmi_prof_* (bench.py's roofline leg reads it).
#include "common.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
thread_local char g_err[512] = "";

struct ProfRec {
    int cls;
    hipEvent_t start, stop;
    double flops, bytes;
};
bool g_prof_on = false;
std::mutex g_prof_mu;
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

void apexmi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int apexmi_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        apexmi_set_error("%s: %s", what, hipGetErrorString(e));
        return 2;
    }
    return 0;
}

ApexmiProfScope::ApexmiProfScope(int cls_, hipStream_t s, double flops, double bytes)
    : cls(cls_), stream(s), slot(-1) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    r.cls = cls;
    r.start = get_event();
    r.stop = get_event();
    r.flops = flops;
    r.bytes = bytes;
    (void)hipEventRecord(r.start, stream);
    slot = (int)g_recs.size();
    g_recs.push_back(r);
}

ApexmiProfScope::~ApexmiProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_recs[slot].stop, stream);
}

extern "C" int apexmi_version(void) { return 100; }  // 0.1.0

extern "C" const char* apexmi_last_error(void) { return g_err; }

extern "C" int apexmi_prof_enable(int on) {
    g_prof_on = on != 0;
    return 0;
}

extern "C" int apexmi_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_recs) {
        g_pool.push_back(r.start);
        g_pool.push_back(r.stop);
    }
    g_recs.clear();
    return 0;
}

extern "C" int apexmi_prof_read(double ms[APEXMI_NCLASS], int64_t launches[APEXMI_NCLASS],
                                double flops[APEXMI_NCLASS], double bytes[APEXMI_NCLASS]) {
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        apexmi_set_error("prof_read: %s", hipGetErrorString(e));
        return 2;
    }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int c = 0; c < APEXMI_NCLASS; ++c) {
        ms[c] = 0;
        launches[c] = 0;
        flops[c] = 0;
        bytes[c] = 0;
    }
    for (auto& r : g_recs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.start, r.stop) != hipSuccess) continue;
        ms[r.cls] += t;
        launches[r.cls] += 1;
        flops[r.cls] += r.flops;
        bytes[r.cls] += r.bytes;
    }
    return 0;
}

// ---- tuning switches (A/B levers; defaults are the shipped configuration, include/apexmi.h lists the keys) ----
void apexmi_set_attn_waves(int v);
void apexmi_set_attn_mfma(int v);
void apexmi_set_ln_wave(int v);
void apexmi_set_attn_c4(int v);
void apexmi_set_attn_stages(int v);
void apexmi_set_qk_group(int v);
void apexmi_set_attn_split(int v);
void apexmi_set_conv_v2(int v);
void apexmi_set_conv_slab(int v);
void apexmi_set_conv_pp(int v);
void apexmi_set_conv_dbg(int v);
void apexmi_set_conv_torder(int v);
void apexmi_set_conv_prof(int half, int v);
int apexmi_set_gemm_key(const char* key, int value);

extern "C" int apexmi_tune_set(const char* key, int value) {
    if (!key) {
        apexmi_set_error("tune_set: null key");
        return 1;
    }
    if (!strncmp(key, "gemm.", 5)) {
        if (apexmi_set_gemm_key(key, value) == 0) return 0;
    } else if (!strcmp(key, "conv.v2")) {
        apexmi_set_conv_v2(value);
        return 0;
    } else if (!strcmp(key, "conv.slab")) {
        apexmi_set_conv_slab(value);
        return 0;
    } else if (!strcmp(key, "conv.prof_lo") || !strcmp(key, "conv.prof_hi")) {
        apexmi_set_conv_prof(key[10] == 'h', value);
        return 0;
    } else if (!strcmp(key, "conv.torder")) {
        apexmi_set_conv_torder(value);
        return 0;
    } else if (!strcmp(key, "conv.dbg")) {
        apexmi_set_conv_dbg(value);
        return 0;
    } else if (!strcmp(key, "conv.pp")) {
        apexmi_set_conv_pp(value);
        return 0;
    } else if (!strcmp(key, "attn.split")) {
        apexmi_set_attn_split(value);
        return 0;
    } else if (!strcmp(key, "qk.group")) {
        apexmi_set_qk_group(value);
        return 0;
    } else if (!strcmp(key, "attn.stages")) {
        apexmi_set_attn_stages(value);
        return 0;
    } else if (!strcmp(key, "attn.waves")) {
        apexmi_set_attn_waves(value);
        return 0;
    } else if (!strcmp(key, "attn.c4")) {
        apexmi_set_attn_c4(value);
        return 0;
    } else if (!strcmp(key, "ln.wave")) {
        apexmi_set_ln_wave(value);
        return 0;
    } else if (!strcmp(key, "attn.mfma")) {
        apexmi_set_attn_mfma(value);
        return 0;
    }
    apexmi_set_error("tune_set: unknown key");
    return 1;
}
